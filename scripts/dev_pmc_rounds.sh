for cfg in "8192 512" "16384 512" "65536 512" "65536 2560"; do set -- $cfg
  export DEV_M=$1
  PMC_GROUPS="FETCH_SIZE;TCC_HIT_sum TCC_MISS_sum" scripts/prof_pmc.sh gpurun_out/pmc_round_$1_$2.txt z 65536 $2 3 > /dev/null 2>&1
  echo "M=$1 n=$2: $(grep -E "FETCH_SIZE|TCC_HIT_sum|TCC_MISS_sum" gpurun_out/pmc_round_$1_$2.txt | grep -v group | awk '{print $1, $3}' | tr '\n' ' ') | $(grep HEMM /tmp/pmc_0.log | tail -1 | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}')"
done
