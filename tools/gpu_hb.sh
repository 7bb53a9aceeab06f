cd $GRAFT_REPO_ROOT
cp eppm_amd/lib/libeppm_hip.so /tmp/orig.so
for v in old cur old cur; do cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so; echo == $v; python tools/host_boundary.py; done
cp /tmp/orig.so eppm_amd/lib/libeppm_hip.so
