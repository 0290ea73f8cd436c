#!/bin/bash
# sparse Gram matrix from the nonzeros against the dense MFMA Gram, rows of 4 .. 20 nonzeros, banded and scattered
mkdir -p gpurun_out
{
for k in 4 8 9 12 16 20; do python tools/time_spgram.py 1e7 128 $k; done
for k in 12 16; do python tools/time_spgram.py 1e7 128 $k --scatter; done
python tools/time_spgram.py 5e6 512 12
python tools/time_spgram.py 5e6 512 16
python -m pytest tests/test_sparse.py -m gpu -x -q 2>&1 | tail -2
} > gpurun_out/spgram.txt 2>&1
cat gpurun_out/spgram.txt
