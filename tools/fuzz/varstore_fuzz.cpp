// ASan/UBSan fuzz of the VarStore reader: byte flips, truncations, splices of a valid archive; every outcome must be a blob or an ocr::Error.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>
#include "common.hpp"
int main(int argc, char** argv) {
  std::ifstream in(argv[1], std::ios::binary);
  std::vector<unsigned char> base((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  std::mt19937 rng(1234);
  int ok = 0, err = 0;
  const std::string tmp = std::string(argc > 3 ? argv[3] : "/tmp") + "/varstore_fuzz_case.ot";
  for (int it = 0; it < iters; ++it) {
    std::vector<unsigned char> f = base;
    const int kind = rng() % 5;
    if (kind == 0) { const int k = 1 + rng() % 4; for (int j = 0; j < k; ++j) f[rng() % f.size()] ^= (unsigned char)(1u << (rng() % 8)); }
    else if (kind == 1) { const int k = 1 + rng() % 8; for (int j = 0; j < k; ++j) f[rng() % f.size()] = (unsigned char)rng(); }
    else if (kind == 2) f.resize(rng() % f.size());
    else if (kind == 3) { const size_t a = rng() % f.size(), n = 1 + rng() % 64; for (size_t j = a; j < a + n && j < f.size(); ++j) f[j] = 0xff; }
    else { const size_t a = rng() % f.size(), b = rng() % f.size(), n = 1 + rng() % 128; for (size_t j = 0; j < n && a + j < f.size() && b + j < f.size(); ++j) f[a + j] = base[b + j]; }
    { std::ofstream o(tmp, std::ios::binary | std::ios::trunc); o.write((const char*)f.data(), (std::streamsize)f.size()); }
    for (int k = 1; k <= 2; ++k) {
      try { auto blob = ocr::varstore_to_blob(tmp.c_str(), k); ++ok; (void)blob; }
      catch (const ocr::Error&) { ++err; }
      catch (const std::bad_alloc&) { ++err; printf("bad_alloc at iteration %d\n", it); }
      catch (const std::length_error&) { ++err; printf("length_error at iteration %d\n", it); }
    }
  }
  printf("%d parsed, %d refused\n", ok, err);
  return 0;
}
