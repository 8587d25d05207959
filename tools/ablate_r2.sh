# kernel times of ablated library variants (MNMF_ABLATE builds copied to manta_amd/libmnmf_ab<N>.so), streams one after the other
# usage: tools/ablate_r2.sh "<suffixes>"   e.g. "_ab5 _ab6"
export MNMF_XT_OVERLAP=0
for a in $1; do
  echo "== lib$a"
  MNMF_LIB=$GRAFT_REPO_ROOT/manta_amd/libmnmf$a.so bash $GRAFT_REPO_ROOT/tools/prof_configs.sh abl$a "c2" || exit 1
done
