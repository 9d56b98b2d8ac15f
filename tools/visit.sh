bash tools/final_r05.sh
