for v in base nontE; do for b in 2 4 8 16 32; do tools/abv.sh ocean1024 $b 3200 $v; done; done
