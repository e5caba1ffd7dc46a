for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline --opt gate_bits=1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bits on ', d['ms_per_step'], d['roofline']['frac'])"
  timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline --opt gate_bits=0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bits off', d['ms_per_step'], d['roofline']['frac'])"
done
