for p in 2 3 4; do for lib in default lb23; do
  echo "== CTL_PERSIST=$p lib=$lib"
  if [ "$lib" = default ]; then unset CTL_HIP_LIB; else export CTL_HIP_LIB=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_$lib.so; fi
  CTL_PERSIST=$p python tools/bench_conv.py child fwd 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:]); print('  '+'  '.join('%s %s'%(k,v[0]) for k,v in d.items()))
"
done; done
