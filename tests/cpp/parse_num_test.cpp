// Reads one token per line from stdin, prints "<consumed> <flags> <bits of the double as hex>" per line
// (parse_float) or, with argument "int", "<consumed> <value>" (parse_int).  Driven by tests/test_parse_num.py.
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>

#include "../../nimfm_amd/csrc/parse_num.h"

static const nfm::num::Pow5 kTable[] = NFM_POW5_TABLE_INIT;

int main(int argc, char** argv) {
  const bool ints = argc > 1 && !strcmp(argv[1], "int");
  std::string line;
  while (std::getline(std::cin, line)) {
    if (ints) {
      int64_t v = 0;
      const int c = nfm::num::parse_int(line.data(), (int64_t)line.size(), &v);
      printf("%d %lld\n", c, (long long)v);
    } else {
      double v = 0.0;
      int flags = 0;
      const int c = nfm::num::parse_float(line.data(), (int64_t)line.size(), &v, &flags, kTable);
      uint64_t b;
      memcpy(&b, &v, 8);
      printf("%d %d %016llx\n", c, flags, (unsigned long long)b);
    }
  }
  return 0;
}
