set -x
bash tools/profile_round.sh r06 3 3 2>&1 | tail -15
bash tools/profile_round.sh r06 2 5 2>&1 | tail -15
