cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3sweep2; S=gpurun_out/r3sweep2/sweep_factors_24_26_colsw.jsonl; : > $S
timeout -k 10 200 python3 tools/sweep.py --lg 24 --batch 1 --reps 30 --set "" --set "factors=9.7.8,colsw=1" --set "factors=8.8.8,colsw=1" --set "factors=8.7.9,colsw=1" >> $S 2>&1 && \
timeout -k 10 200 python3 tools/sweep.py --lg 24 --batch 16 --reps 9 --set "" --set "factors=9.7.8,colsw=1" --set "factors=8.8.8,colsw=1" --set "factors=8.7.9,colsw=1" --set "factors=9.6.9,colsw=1" >> $S 2>&1 && \
timeout -k 10 200 python3 tools/sweep.py --lg 25 --batch 128 --reps 5 --set "" --set "factors=9.8.8,colsw=1" --set "factors=8.8.9,colsw=1" --set "factors=9.7.9,colsw=1" --set "factors=8.9.8,colsw=1" >> $S 2>&1 && \
timeout -k 10 200 python3 tools/sweep.py --lg 26 --batch 64 --reps 5 --set "" --set "factors=9.8.9,colsw=1" --set "factors=8.9.9,colsw=1" --set "factors=9.9.8,colsw=1" --set "factors=8.8.10,colsw=1" >> $S 2>&1 && \
timeout -k 10 200 python3 tools/sweep.py --lg 28 --batch 16 --reps 5 --set "" --set "factors=9.9.10,colsw=1" --set "factors=8.10.10,colsw=1" --set "factors=9.10.9,colsw=1" >> $S 2>&1
echo rc=$?
