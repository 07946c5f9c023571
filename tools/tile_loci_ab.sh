for tl in 64 0 59 56 60 62 64 0; do echo "tile_loci $tl: $(PISCES_HIP_TILE_LOCI=$tl python tools/chain_bench.py --reps 10 | tail -1 | sed 's/.*device chain//')"; done
python tools/pair_bench.py | grep pair_bench
