cd /root/repo
bash exp_libs/run.sh dppmm gogdpp
cp exp_libs/gogdpp.so plasticinelab_amd/libplmpm.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
