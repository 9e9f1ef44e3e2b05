for f in 3 8 32; do
MI355R_LANES=1 MI355R_BATCH=1 timeout -k 10 120 python bench.py --steps 2 --warmup 1 --frames $f --no-cpu > gpurun_out/diag$f.out 2> gpurun_out/diag$f.err
rc=$?
echo frames=$f rc=$rc
grep -a -i "mi355r\|fault\|stage" gpurun_out/diag$f.err | head -20
if [ $rc -ne 0 ]; then exit 1; fi
done
