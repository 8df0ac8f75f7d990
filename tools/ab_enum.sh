# A/B of the burst generator: tools/ab_enum.sh "VARIANT ..."  (VARIANT = base | a build_ab/libNAME.so name | levels=N)
for v in ${1:-base}; do
  unset THETA_HIP_LIB THETA_ENUM_LEVELS THETA_ENUM_LEGACY THETA_ENUM_PER_TASK
  case $v in
    base) ;;
    levels=*) export THETA_ENUM_LEVELS=${v#levels=} ;;
    legacy) export THETA_ENUM_LEGACY=1 ;;
    pertask=*) export THETA_ENUM_PER_TASK=${v#pertask=} ;;
    *) export THETA_HIP_LIB=$PWD/build_ab/lib$v.so ;;
  esac
  echo "== $v"; timeout 120 python tools/enum_burst_probe.py $2 $3 2>&1 | grep "2^28"
done
