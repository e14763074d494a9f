python scripts/debug/epoch_gap.py 2>&1 | grep "profile True"
IRSPACK_AMD_LIB=$GRAFT_REPO_ROOT/irspack_amd/variants/libirspack_amd_occ16.so python scripts/debug/epoch_gap.py 2>&1 | grep "profile True"
