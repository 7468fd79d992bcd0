export SRF_COMMIT=699b738
bash tools/profile_round.sh r04_m > gpurun_out/r04_m_profile_round.log 2>&1
tail -2 gpurun_out/r04_m_profile_round.log | cut -c1-300
