set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03prof_conv
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -o p -- python3 $R/tools/cfg4_cnn_grad_profile.py > /dev/null 2> $O/pmc_$tag.err
  echo "== --pmc $c" >> $O/pmc_conv_summary.txt
  python3 $R/tools/pmc_summary.py $O/pmc_$tag conv_ >> $O/pmc_conv_summary.txt
  rm -rf $O/pmc_$tag
done
echo done
