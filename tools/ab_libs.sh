# A/B of kernel builds on the per-layer microbenchmark: tools/ab_libs.sh fwd|wgrad lib1 lib2 ...
mode=$1; shift
for lib in "$@"; do
  echo "== $lib"
  if [ "$lib" = default ]; then unset CTL_HIP_LIB; else export CTL_HIP_LIB=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_$lib.so; fi
  python tools/bench_conv.py child $mode 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:])
        for k,v in d.items(): print('  %-28s %s'%(k,v))
"
done
