cd /root/repo
bash exp_libs/run.sh lossfk sort32
cp exp_libs/sort32.so plasticinelab_amd/libplmpm.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
