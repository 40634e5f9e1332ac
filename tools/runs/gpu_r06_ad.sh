cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_gpu_vcf.py -q -n 6 -k "piped or gzip_stream" 2>&1 | tail -3
