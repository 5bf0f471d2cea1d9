export TMPDIR=/tmp
OUT=gpurun_out/r3l
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
F=$(find $OUT/pmc_fetch -name "f_counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "w_counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W cnn+solver 1024 > $OUT/pmc_hbm_traffic.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
python3 -c "
import json; d=json.load(open('$OUT/pmc_hbm_traffic.json'))
for k,v in d.items():
    if not k.startswith('_'): print(k[:50], v)
"
