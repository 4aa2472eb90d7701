# round 4, call 41: texture maps before the state machine in EVERY instantiation of the interpreter (mapsbefore: out of line; mapsall: inline) against the shipped choice (mode 4 only, inline)
bash profiles/variants.sh "mapsbefore mapsall" aquarium "aquarium --traversal hier" water-glass "water-glass --traversal hier" "aquarium --samples 64 --steps 2 --traversal hier" > gpurun_out/c41_variants.txt 2>&1
cat gpurun_out/c41_variants.txt
