cd /root/repo
bash tools/r04_final_prof.sh $1 2>&1 | tail -45
