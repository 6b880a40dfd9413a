set -x
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu -k "tuned_workspace or two_ranks_on_one_gpu or one_launch_adam or one_call or forward_backward" 2>&1 | tail -15
for G in 1 2 4 7; do SKS_BIN_GROUPS=$G python tools/bench_stress_forms.py 5 2>&1 | tail -1; done
SKS_BIN_GROUPS=4 python tools/bench_stress_forms.py 5 -1 2>&1 | tail -1
SKS_BIN_GROUPS=2 python tools/bench_stress_forms.py 5 -1 2>&1 | tail -1
