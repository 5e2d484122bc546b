cd $GRAFT_REPO_ROOT
for br in 8 4 2 16; do echo "block_rows $br"; SHARD_BLOCK_ROWS=$br timeout -k 10 200 python tools/shard_perf.py 1920 512 8 2>&1 | grep "G="; done
