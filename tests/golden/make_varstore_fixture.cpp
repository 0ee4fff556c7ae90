// Writes weight archives produced by the same libtorch C++ calls that tch 0.3.0's VarStore::save reaches
// (torch-sys at_save_multi: OutputArchive::write(name, tensor) per variable, then save_to).  Data generator for the
// importer tests, not product code.
//   T=$(python -c "import torch,os;print(os.path.dirname(torch.__file__))")
//   g++ -std=c++17 -D_GLIBCXX_USE_CXX11_ABI=1 -I$T/include -I$T/include/torch/csrc/api/include \
//       make_varstore_fixture.cpp -o /tmp/mkvs -L$T/lib -ltorch -ltorch_cpu -lc10 -Wl,-rpath,$T/lib
//   /tmp/mkvs tests/golden/varstore_small.ot              the committed small fixture
//   /tmp/mkvs out.ot specs.txt                            one tensor per line of specs.txt: "<name> <d0> [<d1> ...]",
//                                                         element i of tensor t = ((i % 251) - 125) / 128 + t  (exact in f32)
#include <ATen/ATen.h>
#include <torch/serialize/archive.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  torch::serialize::OutputArchive archive;
  if (argc >= 3) {
    std::ifstream in(argv[2]);
    std::string line;
    int t = 0;
    while (std::getline(in, line)) {
      std::istringstream ls(line);
      std::string name;
      if (!(ls >> name)) continue;
      std::vector<int64_t> shape;
      int64_t d, count = 1;
      while (ls >> d) {
        shape.push_back(d);
        count *= d;
      }
      at::Tensor idx = at::arange(count, at::kLong);
      at::Tensor v = (idx.remainder(251) - 125).to(at::kFloat) / 128.0 + (double)t;
      archive.write(name, v.reshape(shape), /*is_buffer=*/false);
      ++t;
    }
    archive.save_to(argv[1]);
    return 0;
  }
  struct Spec { const char* name; std::vector<int64_t> shape; double scale; };
  // the recogniser's VarStore (char_recognition/model.rs:13-24) cut down to tiny shapes, plus BN-style
  // statistics as the detector stores them (running stats are plain variables in a tch VarStore)
  const std::vector<Spec> specs = {
      {"conv1.weight", {4, 1, 5, 5}, 0.01},       {"conv1.bias", {4}, 0.1},
      {"fc2.weight", {3, 8}, -0.02},              {"fc2.bias", {3}, 1.0},
      {"layer1.0.bn1.running_mean", {4}, 0.25},   {"layer1.0.bn1.running_var", {4}, 2.0},
      {"layer2.0.downsample.1.weight", {2}, 3.0},
  };
  for (const Spec& s : specs) {
    int64_t count = 1;
    for (int64_t d : s.shape) count *= d;
    at::Tensor t = (at::arange(count, at::kFloat) * s.scale + 1.0).reshape(s.shape);  // value i*scale + 1
    archive.write(std::string(s.name), t, /*is_buffer=*/false);
  }
  archive.save_to(argv[1]);
  return 0;
}
