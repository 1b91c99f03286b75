timeout -k 10 500 python bench.py > gpurun_out/bdef.json 2> gpurun_out/bdef.err; tail -c 2500 gpurun_out/bdef.json; tail -3 gpurun_out/bdef.err
