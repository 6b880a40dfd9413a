set -x
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q -m gpu -k "view_groups or tuned_workspace or one_call" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_api_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('value', round(r['value']), 'ms', round(r['ms_per_step'],5), 'frac', round(ro['frac'],4), 'same_buf', round(ro['frac_same_buffer'],4), 'rot_us', round(ro['avg_launch_us'],2), 'zero_fresh', ro.get('zero_fill_fresh_memory_us'), 'one_call', r['one_call_step']['ms_per_step'])"
python tools/fuzz_binned.py 60 7 raster 2>&1 | tail -3
