# tile length vs resident waves for the interpreter; run on the GPU box from the repo root:  bash tools/occ_probe.sh
for t in 4 8 12 16 20 24 28 32; do
  echo "## SRACK_TILE_MAX=$t"
  SRACK_TILE_MAX=$t bash tools/variants.sh "--flags 2" "--flags 6" | grep -v "^==" | cut -c1-40
  SRACK_TILE_MAX=$t timeout 300 python tools/p3_bench.py 131072 | cut -c1-60
done
