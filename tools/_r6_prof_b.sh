set -x
bash tools/profile_round.sh r06 4 2 2>&1 | tail -15
bash tools/profile_round.sh r06 5 1 2>&1 | tail -15
