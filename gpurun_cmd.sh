cd /root/repo
bash exp_libs/run.sh strided fly
cp exp_libs/fly.so plasticinelab_amd/libplmpm.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
