R=$GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['streams'][:40])"; }
echo "base same-stream"; run
echo "base --overlap"; run --overlap
for v in ns4 ns3; do
  echo "$v same-stream"; VQF_LIB=$R/variants/libvqf_f_$v.so run
  echo "$v --overlap"; VQF_LIB=$R/variants/libvqf_f_$v.so run --overlap
  echo "$v --overlap, LSTM products on the 128x128 kernel"; VQF_GEMM_F32_WAVE=0 VQF_LIB=$R/variants/libvqf_f_$v.so run --overlap
done
