// write_ply.cpp -- produces the reference's PLY export byte stream with the reference's own vendored tinyply
// (external/tinyply/source/tinyply.h), so that the product's exporter (gs-livm_amd/ply.py + the device row packer)
// can be checked BYTE FOR BYTE against what GS-LIVM writes (src/gs/gaussian.cu: construct_list_of_attributes
// :474-492, Save_ply :494-522, Write_output_ply :542-573).
//
// Built ONLY in the build container, directly from the reference's header-only library where it lies under
// /root/reference (recipe: oracle/Makefile, target _ref/write_ply; output stays in oracle/_ref/, git-ignored).
// It contains no reference source text: it calls tinyply's public API in the sequence the reference's writer
// uses -- one add_properties_to_element("vertex", names, FLOAT32, count, data, INVALID, 0) per tensor, in the
// order xyz, normals, f_dc, f_rest, opacity, scale, rotation, then write(stream, /*binary=*/true).
//
// usage: write_ply <P> <M> <in.f32> <out.ply>
//   in.f32 = the seven row-major f32 arrays back to back: xyz[P][3], normals[P][3], f_dc[P][3],
//   f_rest[P][3(M-1)], opacity[P][1], scale[P][3], rot[P][4]   (f_dc / f_rest already in the reference's
//   "transpose(1,2).flatten(1)" channel-major column order; tests/golden/make_golden_ply.py prepares them)
#define TINYPLY_IMPLEMENTATION
#include <tinyply.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const size_t P = (size_t)atol(argv[1]);
  const int M = atoi(argv[2]);
  const size_t cols[7] = {3, 3, 3, (size_t)(3 * (M - 1)), 1, 3, 4};
  std::vector<std::string> names = {"x", "y", "z", "nx", "ny", "nz"};
  for (int i = 0; i < 3; i++) names.push_back("f_dc_" + std::to_string(i));
  for (int i = 0; i < 3 * (M - 1); i++) names.push_back("f_rest_" + std::to_string(i));
  names.push_back("opacity");
  for (int i = 0; i < 3; i++) names.push_back("scale_" + std::to_string(i));
  for (int i = 0; i < 4; i++) names.push_back("rot_" + std::to_string(i));

  std::vector<std::vector<float>> data(7);
  FILE* f = fopen(argv[3], "rb");
  if (!f) return 3;
  for (int t = 0; t < 7; t++) {
    data[t].resize(P * cols[t]);
    if (cols[t] && fread(data[t].data(), sizeof(float), P * cols[t], f) != P * cols[t]) return 4;
  }
  fclose(f);

  tinyply::PlyFile ply;
  size_t off = 0;
  for (int t = 0; t < 7; t++) {
    std::vector<std::string> cur(names.begin() + off, names.begin() + off + cols[t]);
    ply.add_properties_to_element("vertex", cur, tinyply::Type::FLOAT32, P,
                                  reinterpret_cast<uint8_t*>(data[t].data()), tinyply::Type::INVALID, 0);
    off += cols[t];
  }
  std::filebuf fb;
  fb.open(argv[4], std::ios::out | std::ios::binary);
  std::ostream os(&fb);
  ply.write(os, true);
  return 0;
}
