for rep in 1 2; do for m in 0 1 2; do
echo "LM_FE_SHARE=$m rep $rep: $(LM_FE_SHARE=$m python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done; done
LM_FE_SHARE=2 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or reference_lines or live_stream" 2>&1 | tail -2
