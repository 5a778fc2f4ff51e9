cd /root/repo
bash exp_libs/run.sh sort32 dppmm
cp exp_libs/dppmm.so plasticinelab_amd/libplmpm.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
