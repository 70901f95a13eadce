export GPU_MAX_HW_QUEUES=16
bash tools/refresh_r04.sh || exit 1
rm -f gpurun_out/final4/r04_host_abi_sizes.txt gpurun_out/final4/r04_in_library_split.txt gpurun_out/final4/r04_conc_final.txt
bash tools/refresh_r04.sh part2 || exit 1
