#!/bin/bash
# usage: scripts/ab.sh ROUNDS SOLVERS name=envOrLib ...   (development: alternate variants on one box)
# each variant is "label:ENV=VAL" or "label:lib=NAME" (irspack_amd/variants/libirspack_amd_NAME.so) or "label:"
R=$1; S=$2; shift 2
for i in $(seq $R); do
  for v in "$@"; do
    label=${v%%:*}; spec=${v#*:}
    (
      if [[ $spec == lib=* ]]; then export IRSPACK_AMD_LIB=$GRAFT_REPO_ROOT/irspack_amd/variants/libirspack_amd_${spec#lib=}.so;
      elif [[ -n $spec ]]; then export "$spec"; fi
      timeout 300 python scripts/quick_ials.py --solvers $S --epochs ${EPOCHS:-9} ${QARGS} 2>&1 | grep '"solver"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); k=d['kernels']
    print('$label', d['solver'], 'epoch', d['median_ms'], {n.replace('ials_',''):v['ms_per_launch'] for n,v in k.items() if 'solve' in n or 'short' in n})
"
    )
  done
done
