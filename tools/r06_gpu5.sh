for rep in 1 2 3; do
for L in shipped requnm; do
if [ $L = shipped ]; then unset SKS_LIB_OVERRIDE; else export SKS_LIB_OVERRIDE=$PWD/skelsplat_amd/ab_$L.so; fi
python tools/bench_stress.py 2>&1 | grep -E "^backward|^fwd\+bwd" | tr '\n' ' ' | sed "s/^/$L: /"; echo
done; done
unset SKS_LIB_OVERRIDE
SKS_LIB_OVERRIDE=$PWD/skelsplat_amd/ab_requnm.so python tools/fuzz_binned.py 40 3 raster 2>&1 | tail -1
SKS_LIB_OVERRIDE=$PWD/skelsplat_amd/ab_requnm.so bash tools/pmc_table.sh k_render_bwd_tile tools/bench_stress.py 2>&1 | tail -14
