for fw in 1.0 0.7 0.5; do echo "first_w $fw"; IRSPACK_AMD_KNN_CHUNK_FIRST=$fw timeout 200 python scripts/debug/knn_wall.py 2 3 4 2>/dev/null | grep "^chunks" | cut -c1-130; done
