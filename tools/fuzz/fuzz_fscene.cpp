// Host-only fuzz harness: mutated .fscene documents through vkr_fscene_point_lights under ASan/UBSan.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "vkr_c.h"
int32_t vkr_fail(int32_t code, const char* fmt, ...) { (void)fmt; return code; }
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); std::string base; char buf[4096]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) base.append(buf, n);
  fclose(f);
  const long iters = atol(argv[2]);
  std::vector<vkr_light_info> out(64);
  long ok = 0, bad = 0, oom = 0;
  const char* tokens[] = {"{", "}", "[", "]", ",", ":", "\"", "\\", "null", "true", "1e999", "-", "0.", "\"type\"", "\"point_light\"", "\"lights\"", "\\u00", "\\ud800", "e+", "\x00", "\xff"};
  for (long it = 0; it < iters; ++it) {
    std::string t = base;
    const int muts = 1 + rnd() % 4;
    for (int m = 0; m < muts && !t.empty(); ++m) {
      const size_t pos = rnd() % t.size();
      switch (rnd() % 6) {
        case 0: t[pos] = (char)rnd(); break;
        case 1: t.erase(pos, 1 + rnd() % 8); break;
        case 2: { const char* k = tokens[rnd() % (sizeof tokens / sizeof *tokens)]; t.insert(pos, k, k[0] ? strlen(k) : 1); break; }
        case 3: t.resize(pos); break;
        case 4: t.insert(pos, t.substr(rnd() % t.size(), rnd() % 64)); break;
        case 5: t.insert(pos, std::string(1 + rnd() % 2000, "[{"[rnd() % 2])); break;  // deep nesting
      }
    }
    // exact-size heap copy so that any read past the end is caught
    char* mem = (char*)malloc(t.size() ? t.size() : 1);
    memcpy(mem, t.data(), t.size());
    uint32_t count = 0;
    const uint32_t cap = rnd() % 3 ? 64 : rnd() % 4;
    const int32_t rc = vkr_fscene_point_lights(mem, t.size(), (rnd() % 5) ? out.data() : nullptr, cap, &count);
    free(mem);
    if (rc == VKR_OK) ++ok; else if (rc == VKR_ERR_OUT_OF_MEMORY) ++oom; else ++bad;
  }
  printf("ok %ld rejected %ld capacity %ld\n", ok, bad, oom);
  return 0;
}
