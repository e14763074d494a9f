#!/bin/bash
# runs quick_ials for each variant library given as arguments
for v in "$@"; do
  echo "== variant $v"
  IRSPACK_AMD_LIB=$GRAFT_REPO_ROOT/irspack_amd/variants/libirspack_amd_$v.so python scripts/quick_ials.py --shape ml20m --K 64 --epochs 7 --solvers ${SOLVERS:-CHOLESKY,CG} 2>&1 | grep '"solver"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); k=d['kernels']
    print(d['solver'], 'median_ms', d['median_ms'], 'Mupd/s', round(d['updates_per_s']/1e6,1), {n.replace('ials_',''):v['ms_per_launch'] for n,v in k.items() if 'solve' in n or 'split' in n})
"
done
