for i in 1 2; do
bash profiles/variants.sh "pk_vleaf pk_nomask pk_both packet" "big-scene" "big-scene --traversal hier"
done > gpurun_out/c54_ab.log 2>&1
