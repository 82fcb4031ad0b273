# rocprofv3 evidence for the matcher at 100 k x 100 k (VERDICT r2 "Missing" #4): kernel trace + stats, then the MFMA
# counters in their own passes (counters never together with the trace domains other than --kernel-trace)
export TMPDIR=/tmp; root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
python $root/tools/match_prof.py > $out/r03_match_plain.txt 2>&1; cat $out/r03_match_plain.txt
(cd /tmp && rm -rf /tmp/mt && rocprofv3 --kernel-trace --stats -d /tmp/mt -o m --output-format csv -- python $root/tools/match_prof.py > $out/r03_match_trace.log 2>&1)
cp /tmp/mt/*kernel_stats.csv $out/r03_match_kernel_stats.csv 2>/dev/null || find /tmp/mt -name "*kernel_stats.csv" -exec cp {} $out/r03_match_kernel_stats.csv \;
cat $out/r03_match_kernel_stats.csv | head -5
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  d=/tmp/mp_$i; rm -rf $d
  (cd /tmp && MATCH_REPS=2 rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- python $root/tools/match_prof.py > $out/r03_match_pmc_$i.log 2>&1) || tail -3 $out/r03_match_pmc_$i.log
  i=$((i+1))
done
python $root/tools/pmc_sq.py /tmp/mp_0 /tmp/mp_1 /tmp/mp_2 /tmp/mp_3 /tmp/mp_4 > $out/r03_match_pmc.csv; cat $out/r03_match_pmc.csv
