// tensor_loader.h -- name -> host tensor lookup with shape checks and upload, shared by the fp32 model loaders.
#pragma once
#include <hip/hip_runtime.h>

#include <initializer_list>
#include <map>
#include <string>
#include <vector>

#include "../../include/mia.h"

struct TensorLoader {
  std::vector<void*>* allocs = nullptr;      // every device allocation is appended here (freed by the owner)
  std::map<std::string, const mia_tensor_view*> by_name;
  std::string err;

  void index(const mia_tensor_view* tensors, int n) {
    for (int i = 0; i < n; ++i) if (tensors[i].name) by_name[tensors[i].name] = &tensors[i];
  }
  bool has(const std::string& n) const { return by_name.count(n) != 0; }
  bool f32(const std::string& n, std::vector<float>& out, std::initializer_list<int64_t> shp) {
    auto it = by_name.find(n);
    if (it == by_name.end()) { if (err.empty()) err = "missing tensor '" + n + "'"; return false; }
    const mia_tensor_view* t = it->second;
    if (t->dtype != MIA_F32) { if (err.empty()) err = "tensor '" + n + "' must be float32"; return false; }
    int64_t numel = 1; bool ok = t->ndim == (int)shp.size(); int i = 0;
    for (int64_t s : shp) { if (ok && t->shape[i] != s) ok = false; ++i; }
    for (int k = 0; k < t->ndim; ++k) numel *= t->shape[k];
    if (!ok) { if (err.empty()) err = "tensor '" + n + "' has an unexpected shape"; return false; }
    out.assign((const float*)t->data, (const float*)t->data + numel);
    return true;
  }
  float* up(const std::vector<float>& v) {
    void* p = nullptr;
    if (hipMalloc(&p, v.size() * 4 + 64) != hipSuccess) { if (err.empty()) err = "hipMalloc failed"; return nullptr; }
    allocs->push_back(p);
    if (hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess && err.empty()) err = "hipMemcpy failed";
    return (float*)p;
  }
};
