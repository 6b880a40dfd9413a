set -x
timeout 1200 python -m pytest tests/test_raster_gpu.py tests/test_api_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "two_ranks_on_one_gpu and early_stop" 2>&1 | tail -4
for i in 1 2 3; do
for T in 0 1; do
SKS_BWD_TAIL=$T python bench.py --form two --no-extras --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('TAIL=$T', 'ms_per_step', round(r['ms_per_step'],5), 'value', round(r['value']), 'frac', round(r['roofline']['frac'],4), 'same_buf', round(r['roofline']['frac_same_buffer'],4), 'bwd_us', round(r.get('bwd_kernel_avg_us',0),2), r['config']['autotuned']['fill_role'])"
done; done
