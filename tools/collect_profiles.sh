# copy the outputs of tools/gpu_round.sh (gpurun_out/$1) into profiles/ under this round's names ($2, e.g. r03)
O=gpurun_out/$1; P=profiles; R=$2
cp $O/bench.json $P/${R}_bench_default.json; cp $O/bench_nosplit.json $P/${R}_bench_default_nosplit.json
cp $O/bench_bf16x3.json $P/${R}_bench_default_bf16x3.json
cp $O/bench_c3.json $P/${R}_bench_c3.json; cp $O/bench_c4.json $P/${R}_bench_c4.json; cp $O/bench_c5.json $P/${R}_bench_c5.json
cp $O/bench_b32.json $P/${R}_bench_batch32.json
cp $O/bench_infer_c2.json $P/${R}_bench_infer_c2.json; cp $O/bench_infer_c4.json $P/${R}_bench_infer_c4.json
cp $O/kstats.txt $P/${R}_bench_default_kernel_summary.txt; cp $O/kstats_c2_serial.txt $P/${R}_bench_default_serial_kernel_summary.txt
cp $O/kstats_c3_serial.txt $P/${R}_bench_c3_serial_kernel_summary.txt; cp $O/kstats_c4_serial.txt $P/${R}_bench_c4_serial_kernel_summary.txt
cp $O/prof/bench_kernel_stats.csv $P/${R}_bench_default_kernel_stats.csv
cp $O/gpu_tests.log $P/${R}_gpu_tests.log
cp $O/ops_b16.txt $P/${R}_ops_b16.txt; cp $O/ops_b32.txt $P/${R}_ops_b32_sean.txt
cp $O/ops_bf16_c3.txt $P/${R}_ops_bf16_c3.txt; cp $O/ops_bf16_c4.txt $P/${R}_ops_bf16_c4.txt
cp $O/ops_split_b16.txt $P/${R}_ops_split.txt; cp $O/ops_split_b32.txt $P/${R}_ops_split_b32.txt
cp $O/ops_conv9_b16.txt $P/${R}_ops_conv9.txt; cp $O/ops_amax_b16.txt $P/${R}_ops_amax.txt
cp $O/sean_split.json $P/${R}_sean_fwd_dispatch_split.json
