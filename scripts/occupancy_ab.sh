for pad in 0 20480 40960 81920; do
  BHGPU_LIB_OPT_IN=1 BHGPU_LIB=$PWD/gpu-nbody-simulation_amd/build/libbhgpu_exp.so BH_WALK_LDS_PAD=$pad python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('lds_pad $pad', 'walk %.4f' % j['walk_ms'])"
done
