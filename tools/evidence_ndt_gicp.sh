# kernel-level evidence for BASELINE configs #3 (Ndt) and #5 (NdtWithGicp): kernel stats + FETCH_SIZE / WRITE_SIZE passes
# usage: evidence_ndt_gicp.sh <tag>
set -u
export TMPDIR=/tmp
tag=${1:-r02}
out=$PWD/gpurun_out/${tag}_ndt_gicp
mkdir -p "$out"
for w in ndt gicp; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${w}_trace" -- python tools/${w}_probe.py > "$out/${w}_probe.txt" 2> "$out/${w}_trace.err"
  find "$out/${w}_trace" -name '*kernel_stats.csv' -exec cp {} "$out/${w}_kernel_stats.csv" \;
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/${w}_fetch" -- python tools/${w}_probe.py > /dev/null 2> "$out/${w}_fetch.err"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/${w}_write" -- python tools/${w}_probe.py > /dev/null 2> "$out/${w}_write.err"
  python tools/pmc_summary.py "$out/${w}_fetch" ndt_ gicp_ nn_ring grid_ avg_ fitness > "$out/${w}_pmc_fetch_summary.txt"
  python tools/pmc_summary.py "$out/${w}_write" ndt_ gicp_ nn_ring grid_ avg_ fitness > "$out/${w}_pmc_write_summary.txt"
  rm -rf "$out/${w}_trace" "$out/${w}_fetch" "$out/${w}_write"
  cat "$out/${w}_probe.txt"; head -12 "$out/${w}_kernel_stats.csv" | cut -c1-150
  cat "$out/${w}_pmc_fetch_summary.txt" | head -12
done
