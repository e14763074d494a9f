for S in 256 512 1024; do for K in 64 256; do echo "S=$S K=$K"; IRSPACK_AMD_EVAL_SAMPLE=$S IRSPACK_AMD_EVAL_DEBUG=1 python scripts/quick_eval_fused.py $K 2>&1 | tail -2; done; done
