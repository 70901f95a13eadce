bash tools/ab_reduce_rcp.sh || exit 1
bash tools/ab_g2_limb.sh "$@"
