"""Generated-code check for the one kernel that prefetches into REGISTERS from inline asm (conv3x3_c3_fwd_mfma_kernel, csrc/thin.hip).

The next block's 16 patch values are requested by `buffer_load_dword` asm statements and retired by one counted `s_waitcnt` asm; the
compiler does not know the loads are asynchronous.  If register allocation ever puts a copy (or any other read) of a destination
register between its load and the wait, the kernel silently computes on the previous block's patch -- which only shows at sizes with
more than one iteration per wave.  This test cross-compiles the file to gfx950 assembly (no GPU needed) and checks, for every
instance, that inside the main loop no instruction reads a load's destination register between that load and the loop's wait."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_c3_forward_prefetch_registers_are_not_read_before_the_wait(tmp_path):
    src = os.path.join(ROOT, "weather-unet_amd", "csrc", "thin.hip")
    out = str(tmp_path / "thin.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True, timeout=900)
    text = open(out).read().splitlines()
    starts = [i for i, l in enumerate(text) if re.match(r"^_ZN.*conv3x3_c3_fwd_mfma_kernel[^.]*:", l)]       # code labels, not the .kd descriptors
    assert len(starts) >= 8
    checked = 0
    for st in starts:
        end = next(i for i in range(st, len(text)) if "s_endpgm" in text[i])
        body = text[st:end]
        loads = [(i, re.search(r"buffer_load_dword (v\d+),", l).group(1)) for i, l in enumerate(body) if "buffer_load_dword v" in l and "offen" in l]
        # the in-loop group: the LAST 16 asm loads of the function; the loop's wait: the first counted / full vmcnt wait after them
        assert len(loads) >= 32, "expected the pre-loop and the in-loop patch gathers"
        group = loads[-16:]
        first, last = group[0][0], group[-1][0]
        # the loop around them: every label annotated "Loop Header" / "in Loop" next to the loads; it ends at the first label after
        # the loads that is not part of it.  The block with the wait may be laid out BEFORE the header (rotated loop): then the path
        # load -> wait runs to the end of the loop and continues at its first block.
        labels = [(i, l) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
        in_loop = [i for i, l in labels if "Loop" in l]
        loop_start = max([i for i in in_loop if i < first and all("Loop" in l for j, l in labels if i <= j < first)] or [first], key=lambda i: -i)
        after = [i for i, l in labels if i > last and "Loop" not in l]
        loop_end = after[0] if after else len(body)
        is_wait = lambda i: re.search(r"s_waitcnt vmcnt\(\d+\)", body[i]) is not None
        fwd = [i for i in range(last, loop_end) if is_wait(i)]
        if fwd:
            wait, path = fwd[0], lambda at: list(range(at + 1, fwd[0]))
        else:
            early = [i for i in range(loop_start, first) if is_wait(i)]
            assert early, "no vmcnt wait on the loop path after the patch loads"
            wait, path = early[0], lambda at: list(range(at + 1, loop_end)) + list(range(loop_start, early[0]))
        dests = {}
        for i, reg in group:
            dests[reg] = i
        assert len(dests) == 16, "two loads share a destination register"
        for reg, at in dests.items():
            n = int(reg[1:])
            for i in path(at):
                line = body[i].split(";")[0]
                m = re.match(r"\s*(\S+)\s+(.*)", line)
                if not m or m.group(1).startswith(".") or m.group(1).endswith(":"):
                    continue
                ops = [o.strip() for o in m.group(2).split(",")]
                for o in ops[1:]:                     # source operands (single registers and ranges v[a:b])
                    rng = re.match(r"v\[(\d+):(\d+)\]", o)
                    if o == reg or (rng and int(rng.group(1)) <= n <= int(rng.group(2))):
                        raise AssertionError(f"{text[st][:70]}...: `{body[i].strip()}` reads {reg} before the wait (load at +{at}, wait at +{wait})")
        checked += 1
    assert checked == len(starts)


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(ROOT, "weather-unet_amd", "lib", "libwu_kernels.so")


def _device_disassembly(tmp_path):
    """Disassembly of every gfx950 code object bundled in the built library (one per source file)."""
    import shutil
    work = tmp_path / "lib"
    work.mkdir()
    so = str(work / "libwu_kernels.so")
    shutil.copy(LIB, so)                                             # --offloading extracts next to its input
    subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, timeout=300)
    objs = sorted(str(p) for p in work.iterdir() if "amdgcn-amd-amdhsa--gfx950" in p.name)
    assert objs, "no gfx950 code object found in libwu_kernels.so"
    out = []
    for o in objs:
        out.append(subprocess.run([OBJDUMP, "-d", o], check=True, capture_output=True, text=True, timeout=600).stdout)
    return objs, "\n".join(out)


@pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(LIB)), reason="needs llvm-objdump and the built library")
def test_library_has_no_packed_fp32_valu_ops(tmp_path):
    """The guard of DESIGN.md 4 (cross-stream hazard): round 4's discriminating run (profiles/r04_hazard.txt) pinned the corruption of the
    tiled image conv beside the stem's MFMA kernel on the PACKED-FP32 VALU instructions -- every form with v_pk_fma_f32 failed 10 of 10
    whatever its LDS-read queue, every form without was exact, fmaf under -packed-fp32-ops with a deep counted-wait queue included.  Any
    kernel of this library can end up beside an MFMA kernel on a second stream (the GAN step overlaps D with G and with the estimator), so
    NO code object may contain the instruction class; checked on the machine code that ships, every kernel."""
    from wu import _build
    assert not _build.is_stale(), "libwu_kernels.so was not built from the sources in the tree (run __graft_entry__.build())"
    objs, text = _device_disassembly(tmp_path)
    assert len(objs) >= 10 and text.count("s_endpgm") > 100          # the whole library was looked at
    bad = [l.strip() for l in text.splitlines() if re.search(r"\bv_pk_(fma|mul|add)_f32\b", l)]
    assert not bad, f"{len(bad)} packed-FP32 VALU instructions in the shipped library, e.g. {bad[:3]}"
    assert "v_mfma_f32_32x32x16_bf16" in text and "buffer_load_dwordx4" in text      # sanity: this IS the device code


@pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(LIB)), reason="needs llvm-objdump and the built library")
def test_no_vmem_instruction_reads_an_sgpr_a_valu_has_just_written(tmp_path):
    """A vector-memory instruction may not read an SGPR (descriptor word, scalar offset) within 5 wait states of a VALU instruction that wrote
    it (v_readlane_b32 / v_readfirstlane_b32).  The compiler pads its own VMEM instructions; it does NOT see into inline asm, and it restores
    spilled SGPRs with v_readlane_b32 right in front of the instruction that needs them -- round 4's gathered-row data-gradient instance read a
    half-restored descriptor in the first load of a group (gate of channels 0-15 wrong, 16-63 right).  The asm blocks of the LDS-DMA kernels carry
    their own s_nop since; this checks every VMEM instruction of the machine code that ships, along straight-line code."""
    from wu import _build
    assert not _build.is_stale(), "libwu_kernels.so was not built from the sources in the tree (run __graft_entry__.build())"
    _, text = _device_disassembly(tmp_path)
    vmem = re.compile(r"^(buffer_|global_|flat_|scratch_|tbuffer_)")
    sreg = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
    age = {}                # sgpr index -> wait states since a VALU wrote it
    bad, n_vmem = [], 0
    for line in text.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if not m:
            if line.rstrip().endswith(":"):        # a label: another path may arrive here -- keep the fall-through history (the tighter one)
                continue
            continue
        op, args = m.group(1), m.group(2)
        if vmem.match(op):
            n_vmem += 1
            for r in sreg.finditer(args):
                regs = [int(r.group(1))] if r.group(1) else list(range(int(r.group(2)), int(r.group(3)) + 1))
                for x in regs:
                    if age.get(x, 99) < 5:
                        bad.append((line.strip(), x, age[x]))
        step = int(args.split()[0], 0) + 1 if op == "s_nop" else 1
        for x in list(age):
            age[x] += step
            if age[x] > 8:
                del age[x]
        if op in ("v_readlane_b32", "v_readfirstlane_b32"):
            d = re.match(r"s(\d+)", args)
            if d:
                age[int(d.group(1))] = 0
        if op == "s_endpgm":
            age.clear()
    assert n_vmem > 1000
    assert not bad, f"{len(bad)} VMEM instructions read a freshly VALU-written SGPR, e.g. {bad[:3]}"
