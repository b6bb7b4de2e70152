cd $GRAFT_REPO_ROOT
for spec in "wgs512:SSDK_TARGET_WGS=512" "wgs448:SSDK_TARGET_WGS=448" "wgs384:SSDK_TARGET_WGS=384" "wgs320:SSDK_TARGET_WGS=320" "wgs576:SSDK_TARGET_WGS=576" "base:"; do
  tag=${spec%%:*}; envs=${spec#*:}
  echo "== $tag"
  env $envs timeout 200 python tools/scan_probe.py 2>&1 | grep -E "^SURVEY|^all equal|^trained" | cut -c1-150
done
