export SRF_COMMIT=e34bb7d
bash tools/profile_round.sh r04_l > gpurun_out/r04_l_profile_round.log 2>&1
tail -3 gpurun_out/r04_l_profile_round.log
