python scripts/debug/pp_time128.py 128 2>&1 | tail -1
for v in PRED P RANK SOLVE; do IRSPACK_AMD_LIB=$GRAFT_REPO_ROOT/irspack_amd/variants/libirspack_amd_skip_$v.so python scripts/debug/pp_time128.py 128 2>&1 | tail -1; done
