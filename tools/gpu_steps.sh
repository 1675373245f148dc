# source me: run <seconds> <cmd...> -- runs one GPU step under its own timeout; a step that hits its limit ends the whole call
# (no further GPU step after a hang), an ordinary non-zero exit (failed tests) does not.
run() {
  local lim=$1; shift
  timeout -k 10 "$lim" "$@"
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "STEP TIMED OUT ($lim s): $*"; exit $rc; fi
  return 0
}
