python -m pytest tests -m gpu -x -q -k "texture or example or TEX or tex" > gpurun_out/c23_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c23_pytest.log
( for park in 0 1; do export PORTRAYER_PARK=$park; echo "== PORTRAYER_PARK=$park"
bash profiles/variants.sh "mapsinline" "aquarium" "aquarium --traversal hier" "mirror" "cows"
done ) > gpurun_out/c23_var.log 2>&1
PORTRAYER_PARK=1 python -m pytest tests -m gpu -x -q > gpurun_out/c23_pytest_park1.log 2>&1; echo "pytest rc $?" >> gpurun_out/c23_pytest_park1.log
