set -e
R=$GRAFT_REPO_ROOT
export CTU_COMMIT=${CTU_COMMIT:-?}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f --output-format csv -- python $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w --output-format csv -- python $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_w.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w 6 gpurun_out/fin_pmc_hbm_traffic.json > gpurun_out/fin_pmc_hbm_traffic_summary.txt 2>&1
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_serial -- python $R/bench.py --serial --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/ks_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_ov -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/ks_ov.log 2>&1
cd $R
python tools/kstats.py gpurun_out/ks_serial 10 70 > gpurun_out/fin_kernel_stats_serial_per_step.txt
python tools/kstats.py gpurun_out/ks_ov 10 70 > gpurun_out/fin_kernel_stats_overlapped_per_step.txt
cp $(ls gpurun_out/ks_serial/*/*kernel_stats.csv | head -1) gpurun_out/fin_kernel_stats_serial.csv
cp $(ls gpurun_out/ks_ov/*/*kernel_stats.csv | head -1) gpurun_out/fin_kernel_stats_overlapped.csv
rm -rf gpurun_out/ks_serial gpurun_out/ks_ov
python tools/cpu_overhead.py > gpurun_out/fin_cpu_overhead.log 2>&1
