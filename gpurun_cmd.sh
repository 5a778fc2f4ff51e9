cd /root/repo
bash exp_libs/run.sh tiles priv
cp exp_libs/priv.so plasticinelab_amd/libplmpm.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
