// Named-region carving of a caller-provided workspace (host-side bookkeeping only).
#pragma once
#include <stdint.h>
#include <string.h>

namespace ope {

// ---- workspace: a list of named float regions, 256-byte aligned --------------------------------------------
struct Region { const char* name; int64_t off; int64_t n; };
constexpr int kMaxRegions = 160;
struct Workspace {
  Region r[kMaxRegions];
  int n = 0;
  int64_t total = 0;  // floats
  int64_t add(const char* name, int64_t nfloats) {
    const int64_t off = total;
    r[n++] = Region{name, off, nfloats};
    total += (nfloats + 63) & ~(int64_t)63;
    return off;
  }
  int64_t find(const char* name, int64_t* nf) const {
    for (int i = 0; i < n; ++i)
      if (strcmp(r[i].name, name) == 0) {
        if (nf) *nf = r[i].n;
        return r[i].off;
      }
    return -1;
  }
};


}  // namespace ope
