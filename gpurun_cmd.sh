cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCP_[A-Z_0-9]+|TA_[A-Z_0-9]+" | sort -u | tr '\n' ' ' | head -c 6000
