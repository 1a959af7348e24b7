// Host-side constants handed to the sumcheck kernels (Montgomery images, packed labels).
#pragma once
#include "sumcheck_kernels.hip.h"

namespace atlas {

inline ScConsts make_consts() {
    ScConsts K;
    const uint32_t two_inv[8] = {0x1ffffffeu, 0x783c14d8u, 0x0c8d1eddu, 0xaf982f6fu,
                                 0xfcfd4f45u, 0x8f5f7492u, 0x3d9cbfacu, 0x1f37631au};
    const uint32_t six_inv[8] = {0x0aaaaaaau, 0x7d695c48u, 0xaed9b4f4u, 0x3a880fcfu,
                                 0xa9a9c517u, 0xda7526dbu, 0x69deea8eu, 0x0a67cbb3u};
    const uint32_t k32[8] = {0x15b8b9dau, 0x93e78865u, 0xb05ea154u, 0x16df2426u,
                             0x302ab839u, 0x1271b743u, 0xec6c226eu, 0x06bc037eu};
    const uint32_t k64[8] = {0x7c5fb586u, 0xb4c6edf9u, 0xbfeb93beu, 0x708c8d50u,
                             0x04f7e0efu, 0x9ffd1de4u, 0x9a392866u, 0x215b02acu};
    for (int i = 0; i < 8; i++) {
        K.two_inv.v[i] = two_inv[i]; K.six_inv.v[i] = six_inv[i];
        K.k32.v[i] = k32[i]; K.k64.v[i] = k64[i];
    }
    const uint64_t b[4] = {0x5f796c6f50696e55ULL, 0x0000006e69676562ULL, 0, 0};  // "UniPoly_begin"
    const uint64_t e[4] = {0x5f796c6f50696e55ULL, 0x0000000000646e65ULL, 0, 0};  // "UniPoly_end"
    for (int i = 0; i < 4; i++) { K.lbl_begin[i] = b[i]; K.lbl_end[i] = e[i]; }
    return K;
}


}  // namespace atlas
