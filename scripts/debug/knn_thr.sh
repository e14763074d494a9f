python -m pytest tests/test_gpu_knn.py tests/test_gpu_user_knn.py -m gpu -x -q 2>&1 | tail -3
for T in 8 16 32 64 128; do echo "threads $T"; IRSPACK_AMD_KNN_THREADS=$T IRSPACK_AMD_KNN_TIMING=1 python scripts/debug/knn_wall.py 2>&1 | tail -7 | grep -E "target pass|order|wall"; done
