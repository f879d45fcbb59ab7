cd tools/mb
for ns in 8 16; do ./decode_bench 1 2048 $ns 1; done
./decode_bench 1 4096 16 1
