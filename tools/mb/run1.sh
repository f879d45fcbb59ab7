cd tools/mb
for ns in 0 8 16; do ./decode_bench 1 2048 $ns 1; done
for ns in 0 16; do ./decode_bench 1 4096 $ns 1; done
./decode_bench 8 2048 0 1
