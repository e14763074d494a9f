echo == fused; python scripts/quick_eval_fused.py 64 2>&1 | grep "^K" | tail -1; python scripts/quick_eval_fused.py 256 2>&1 | grep "^K" | tail -1
echo == unfused; IRSPACK_AMD_EVAL_FUSED=0 python scripts/quick_eval_fused.py 64 2>&1 | grep "^K" | tail -1; IRSPACK_AMD_EVAL_FUSED=0 python scripts/quick_eval_fused.py 256 2>&1 | grep "^K" | tail -1
