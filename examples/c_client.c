/* c_client.c -- the C side of the drop-in boundary, as a maintainer binding libvallex_hip.so from C (or through cgo / JNI /
 * N-API, which all consume exactly this header) would see it.  Plain C99, no HIP or torch headers.
 *
 *   gcc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude examples/c_client.c -Lvall-e-x_amd/csrc -lvallex_hip \
 *       -Wl,-rpath,$PWD/vall-e-x_amd/csrc -o /tmp/c_client && /tmp/c_client
 *
 * Without arguments it prints the ABI version and the layout of the three descriptor structs (tests/test_abi_c.py compares
 * them with the ctypes binding) and checks the struct_size guard; it never needs a GPU.  With `--run` on an MI355X it also
 * creates a 2-layer context, which fails with VX_ESTATE at vx_infer because no weights were loaded -- the error path a C
 * caller sees.  Loading weights and running inference from C is the sequence INTEGRATION.md section 2 shows in ctypes. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "vallex_hip.h"

#define FIELD(S, F) printf("%s.%s %zu %zu\n", #S, #F, offsetof(S, F), sizeof(((S*)0)->F))

int main(int argc, char** argv) {
  printf("abi %d header %d\n", (int)vx_abi_version(), VX_ABI_VERSION);
  printf("sizeof vx_config %zu\nsizeof vx_batch %zu\nsizeof vx_sampling %zu\n", sizeof(vx_config), sizeof(vx_batch), sizeof(vx_sampling));
  FIELD(vx_config, struct_size); FIELD(vx_config, num_layers); FIELD(vx_config, max_batch); FIELD(vx_config, max_text);
  FIELD(vx_config, max_prompt); FIELD(vx_config, max_new); FIELD(vx_config, use_graph); FIELD(vx_config, with_vocos);
  FIELD(vx_config, debug_taps); FIELD(vx_config, with_encodec); FIELD(vx_config, cu_mask); FIELD(vx_config, arith);
  FIELD(vx_batch, struct_size); FIELD(vx_batch, batch); FIELD(vx_batch, text_ids); FIELD(vx_batch, text_lang);
  FIELD(vx_batch, text_stride); FIELD(vx_batch, text_lens); FIELD(vx_batch, prompt_codes); FIELD(vx_batch, prompt_stride);
  FIELD(vx_batch, prompt_lens);
  FIELD(vx_sampling, struct_size); FIELD(vx_sampling, top_k); FIELD(vx_sampling, temperature); FIELD(vx_sampling, uniforms);
  FIELD(vx_sampling, uniforms_steps); FIELD(vx_sampling, seed); FIELD(vx_sampling, force_eos_at); FIELD(vx_sampling, sync_every);
  FIELD(vx_sampling, best_of); FIELD(vx_sampling, length_penalty); FIELD(vx_sampling, return_worst);

  /* ABI guard: a caller compiled against a shorter (older) vx_config is refused before anything touches the GPU */
  vx_config cfg;
  vx_ctx* ctx = NULL;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = (uint32_t)(sizeof cfg - sizeof cfg.arith);      /* the ABI-3 struct, without `arith` */
  cfg.num_layers = 2; cfg.max_batch = 1; cfg.max_text = 8; cfg.max_prompt = 8; cfg.max_new = 8;
  int rc = vx_create(0, &cfg, &ctx);
  printf("short_struct rc %d msg %s\n", rc, vx_last_error(NULL));
  if (rc != VX_EINVAL || ctx != NULL) return 1;

  if (argc > 1 && strcmp(argv[1], "--run") == 0) {
    cfg.struct_size = (uint32_t)sizeof cfg;
    rc = vx_create(0, &cfg, &ctx);
    printf("create rc %d%s%s\n", rc, rc ? " msg " : "", rc ? vx_last_error(NULL) : "");
    if (rc != VX_OK) return 2;
    int32_t ids[2] = {5, 6}, lang[2] = {0, 0}, tl[1] = {2}, pc[8] = {0}, pl[1] = {1}, out_len[1] = {0};
    int64_t out[8 * 8];
    vx_batch b;
    vx_sampling s;
    memset(&b, 0, sizeof b);
    memset(&s, 0, sizeof s);
    b.struct_size = (uint32_t)sizeof b; b.batch = 1; b.text_ids = ids; b.text_lang = lang; b.text_stride = 2; b.text_lens = tl;
    b.prompt_codes = pc; b.prompt_stride = 1; b.prompt_lens = pl;
    s.struct_size = (uint32_t)sizeof s; s.top_k = 1; s.temperature = 1.0f; s.force_eos_at = -1; s.sync_every = 8; s.best_of = 1;
    s.length_penalty = 1.0f;
    rc = vx_infer(ctx, &b, &s, out, 8, out_len);
    printf("infer without weights rc %d msg %s\n", rc, vx_last_error(ctx));
    vx_destroy(ctx);
    if (rc != VX_ESTATE) return 3;
  }
  return 0;
}
