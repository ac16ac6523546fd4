// check_glm.cpp -- pins the ORACLE's small-matrix helpers to the GLM that the reference vendors
// (external/glm) and computes its covariances with (forward.cu:100-126,164-167; backward.cu:159-200).
//
// Built ONLY in the build container, directly from the reference's own header-only GLM where it lies under
// /root/reference (recipe: oracle/Makefile, target _ref/check_glm; output stays in oracle/_ref/, git-ignored).
// It contains no reference source text: it calls GLM's public API on pseudo-random inputs and dumps the raw
// float results; tests/test_oracle.py compares them bit for bit with gsr_oracle.c's m3_mul / m3_t / dot / length
// restatements (exported for that purpose as gsro_test_*).  What this pins: the accumulation ORDER of
// mat3*mat3, vec3/vec4 dot products and length under g++ -ffp-contract=off -- the only third-party arithmetic
// the exact-match stages (radii, tile rects) depend on.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <glm/glm.hpp>

static uint32_t lcg_state = 12345u;
static float rnd() {  // uniform in [-2, 2), exactly reproducible
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return ((lcg_state >> 8) * (1.0f / 16777216.0f)) * 4.0f - 2.0f;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000;
  for (int it = 0; it < n; it++) {
    glm::mat3 A, B;
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) A[c][r] = rnd();
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) B[c][r] = rnd();
    const glm::mat3 P = A * B;
    const glm::mat3 Q = glm::transpose(A) * glm::transpose(B) * A;  // the shape of cov = T^T Vrk^T T
    glm::vec3 u, v;  // filled component by component: argument evaluation order is unspecified in C++
    glm::vec4 p, q;
    for (int k = 0; k < 3; k++) u[k] = rnd();
    for (int k = 0; k < 3; k++) v[k] = rnd();
    for (int k = 0; k < 4; k++) p[k] = rnd();
    for (int k = 0; k < 4; k++) q[k] = rnd();
    float out[9 + 9 + 4];
    int k = 0;
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) out[k++] = P[c][r];
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) out[k++] = Q[c][r];
    out[k++] = glm::dot(u, v);
    out[k++] = glm::dot(p, q);
    out[k++] = glm::length(u);
    const glm::vec3 d = u / glm::length(u);
    out[k++] = d.x + d.y + d.z;
    fwrite(out, sizeof(float), k, stdout);
  }
  return 0;
}
