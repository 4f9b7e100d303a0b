// wk_tok_internal.h — what the device tokenizer (woltka_hip.hip) needs of the
// host tokenizer object (wk_tokenize.cpp).  Not part of the public ABI.
#pragma once
#include <stdint.h>

#include "../../include/woltka_hip.h"

extern "C" {
int wkx_tok_device_ok(const wk_tok* t);
int32_t wkx_tok_n_names(const wk_tok* t);
void wkx_tok_name(const wk_tok* t, int32_t id, const char** p, uint32_t* len, uint64_t* hash);
int32_t wkx_tok_intern(wk_tok* t, const char* p, uint32_t len);
}
