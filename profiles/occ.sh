for l in 0 60000 100000; do echo "LDS_MIN=$l"; GSD_CONV_LDS_MIN=$l bash profiles/quick_bench.sh --steps 3 --warmup 1; done
