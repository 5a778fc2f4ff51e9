cd /root/repo
bash exp_libs/run.sh shflx tiles
cp exp_libs/phases.so plasticinelab_amd/libplmpm.so
python exp_libs/phases.py 2>&1 | tail -22
cp exp_libs/tiles.so plasticinelab_amd/libplmpm.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
