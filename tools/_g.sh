mkdir -p gpurun_out
timeout -k 10 700 python tools/fuzz.py 1500 51 > gpurun_out/r03_fuzz.txt 2>&1; echo "rc=$?" >> gpurun_out/r03_fuzz.txt
tail -3 gpurun_out/r03_fuzz.txt
timeout -k 10 400 python tools/soak.py 66000 800 > gpurun_out/r03_soak.txt 2>&1; echo "rc=$?" >> gpurun_out/r03_soak.txt
tail -14 gpurun_out/r03_soak.txt
