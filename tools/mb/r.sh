cd tools/mb
for unr in 4 8; do for ns in 4 8; do SPATTEN_DECODE_UNR=$unr ./decode_bench 1 2048 $ns 1; done; done
for unr in 4 8; do for ns in 8 16; do SPATTEN_DECODE_UNR=$unr ./decode_bench 1 4096 $ns 1; done; done
