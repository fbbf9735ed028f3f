# A/B of the camera-block assembly on C2: landmark-major (BSGPU_PAIRS_LM=1) against the camera-pair segments (0)
for e in 1 0; do echo "PAIRS_LM=$e"; BSGPU_PAIRS_LM=$e python scripts/lmc_exp.py 2>&1 | tail -3; done
