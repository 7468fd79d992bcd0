mkdir -p gpurun_out
echo ----- source w5 under rocgdb
( timeout 600 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop nopass" -ex run -ex bt -ex "info threads" --args python bench.py --steps 20 --warmup 5 --loss source --no-cpu-baseline --no-eager-baseline --no-fp32-mode --no-extra-legs --no-roofline 2>&1 | grep -v "^\[New Thread\|^\[Thread\|warning:" | tail -60 | cut -c1-260 )
