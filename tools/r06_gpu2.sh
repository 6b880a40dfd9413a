set -x
python tools/bench_stress_forms.py 5 2>&1 | tail -2
SKS_FUSED_BWD=0 python tools/bench_stress_forms.py 5 2>&1 | tail -2
python tools/bench_stress_forms.py 5 2>&1 | tail -2
timeout 900 python -m pytest tests -x -q -m gpu -k "one_call or forward_backward or two_ranks_on_one_gpu or tuned_workspace" 2>&1 | tail -8
