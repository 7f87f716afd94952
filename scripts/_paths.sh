# Sourced by the scripts: where a binary lives, by what it is -- the stock reference (the checker's: gmix_strict,
# gmix_fast, ref_*) in oracle/_ref/, the reference built WITH the product (dropin/Makefile) in dropin/_build/.
GMX_ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
gmxbin() {
  case "$1" in
    gmix_strict|gmix_fast|ref_tester_strict|ref_trace|ref_mixer_*|ref_indirect_harness|ref_lstm_harness) echo "$GMX_ROOT/oracle/_ref/$1" ;;
    *) echo "$GMX_ROOT/dropin/_build/$1" ;;
  esac
}
