// fold_probe_cli — runs tools/ref_dump/fold_probe.h on one matrix row given as three C99 hexadecimal floats and prints the probe
// (found x y z lo hi, hexadecimal): tests/test_ref_vectors.py checks the construction against the oracle under both folds.
#include <cstdio>
#include <cstdlib>

#include "fold_probe.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const uw_ref_dump::FoldProbe fp = uw_ref_dump::find_fold_probe(std::strtof(argv[1], nullptr), std::strtof(argv[2], nullptr), std::strtof(argv[3], nullptr));
  std::printf("%d %a %a %a %a %a\n", fp.found ? 1 : 0, (double)fp.x, (double)fp.y, (double)fp.z, (double)fp.lo, (double)fp.hi);
  return 0;
}
