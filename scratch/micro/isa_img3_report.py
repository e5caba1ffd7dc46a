"""ISA signatures of the img3 variants (scratch/micro/img3_variants.hip, both builds): per kernel the VGPR count / occupancy, the FP32
FMA forms, the LDS reads, the deepest queue of LDS reads the wave holds behind counted lgkmcnt waits and the wait chain itself.
    python scratch/micro/isa_img3_report.py /tmp/i3/packed.s /tmp/i3/scalar.s"""
import re
import sys


def kernels(txt):
    for m in re.finditer(r"^(_Z[\w]+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        vg = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        yield name, [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", "."))], vg, lds


def report(path):
    txt = open(path).read()
    for name, lines, vg, lds in kernels(txt):
        cnt = lambda *p: sum(l.startswith(p) for l in lines)
        inflight, deepest, waits, smem = 0, 0, [], 0
        for l in lines:
            if l.startswith("ds_read"):
                inflight += 1
                deepest = max(deepest, inflight)
            elif l.startswith(("s_load", "s_buffer_load")):
                inflight += 1
                smem += 1
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if l.startswith("s_waitcnt") and m:
                n = int(m.group(1))
                waits.append(n)
                inflight = min(inflight, n)
        # the longest run of FMAs between two waits = how much arithmetic a counted wait releases
        waves = 512 // ((vg + 7) // 8 * 8) if vg else 8
        print(f"{path.split('/')[-1]:10s} {name}")
        print(f"    VGPRs {vg} (<= {min(waves, 8)} waves/SIMD), LDS {lds} B; v_pk_fma_f32 {cnt('v_pk_fma_f32')}, v_pk_mul/add_f32 {cnt('v_pk_mul_f32', 'v_pk_add_f32')}, "
              f"v_fmac/fma_f32 {cnt('v_fmac_f32', 'v_fma_f32')}, ds_read {cnt('ds_read')} (b128 {cnt('ds_read_b128')}, b64 {cnt('ds_read_b64', 'ds_read2_b32', 'ds_read2_b64')}), "
              f"s_load {smem}")
        print(f"    deepest LDS-read queue {deepest}; lgkmcnt chain {waits}")


for p in sys.argv[1:]:
    report(p)
