p='/root/repo/pienerf_amd/csrc/pn_common.h'
s=open(p).read()
s=s.replace('''    void* fused_levels;       // device, owned: PnFusedLevel[16] (pn_nerf_forward.hip)
''','''    void* wsplit;             // device, owned: bf16 three-way split A-operand image, PN_NET_SPLIT_BYTES (pn_nerf_forward.hip)
    void* fused_levels;       // device, owned: PnFusedLevel[16] (pn_nerf_forward.hip)
''')
s=s.replace('''// internal launcher shared''','''// bf16 image: [20 operand groups][3 splits hi/mid/lo][64 lanes][8 bf16], then the VALU output layer's 192 fp32 weights
#define PN_NET_GROUPS 20
#define PN_NET_SPLIT_W_BYTES (PN_NET_GROUPS * 3 * 64 * 16)
#define PN_NET_SPLIT_BYTES (PN_NET_SPLIT_W_BYTES + 192 * 4)

// internal launcher shared''')
open(p,'w').write(s)

p='/root/repo/pienerf_amd/csrc/pn_nerf_forward.hip'
s=open(p).read()
# host packing
s=s.replace('''    hipError_t e = hipMalloc((void**)&n->wpack, sizeof(float) * PN_NET_MFMAS * 64);
''','''    // bf16 image of the same A operands for v_mfma_f32_32x32x16_bf16: operand group (layer, out tile t, K chunk kc) takes the 8
    // consecutive f32-MFMA slots m0 + 8 kc + e (e = 0..7) of the stream above as the 8 K-elements of its lane, each weight cut
    // into three bf16 pieces w = hi + mid + lo (truncation splits: exact, 8 + 8 + 8 significant bits).
    unsigned char* simg = new unsigned char[PN_NET_SPLIT_BYTES];
    {
        uint16_t* s16 = reinterpret_cast<uint16_t*>(simg);
        int G = 0;
        auto emit = [&](int m0) {
            for (int l = 0; l < 64; l++)
                for (int e2 = 0; e2 < 8; e2++) {
                    float v = host[(m0 + e2) * 64 + l];
                    for (int p = 0; p < 3; p++) {
                        uint32_t u;
                        memcpy(&u, &v, 4);
                        u &= 0xffff0000u;
                        float h;
                        memcpy(&h, &u, 4);
                        s16[((size_t)(G * 3 + p) * 64 + l) * 8 + e2] = (uint16_t)(u >> 16);
                        v -= h;
                    }
                }
            G++;
        };
        for (int t = 0; t < 2; t++) for (int kc = 0; kc < 2; kc++) emit(0 + t * 16 + 8 * kc);    // layer 0: groups 0..3
        for (int kc = 0; kc < 4; kc++) emit(32 + 8 * kc);                                         // layer 1: groups 4..7
        for (int t = 0; t < 2; t++) for (int kc = 0; kc < 2; kc++) emit(64 + t * 16 + 8 * kc);   // layer 2: groups 8..11
        for (int t = 0; t < 2; t++) for (int kc = 0; kc < 4; kc++) emit(96 + t * 32 + 8 * kc);   // layer 3: groups 12..19
        memcpy(simg + PN_NET_SPLIT_W_BYTES, img + PN_NET_VALU_OFF, 192 * 4);
    }
    hipError_t e = hipMalloc((void**)&n->wpack, sizeof(float) * PN_NET_MFMAS * 64);
    if (e == hipSuccess) e = hipMalloc((void**)&n->wsplit, PN_NET_SPLIT_BYTES);
    if (e == hipSuccess) e = hipMemcpyAsync(n->wsplit, simg, PN_NET_SPLIT_BYTES, hipMemcpyHostToDevice, (hipStream_t)stream);
''')
s=s.replace('''    delete[] host;
    delete[] img;
    if (e != hipSuccess) {''','''    delete[] host;
    delete[] img;
    delete[] simg;
    if (e != hipSuccess) {''')
s=s.replace('''    if (n->wpack) (void)hipFree(n->wpack);
''','''    if (n->wpack) (void)hipFree(n->wpack);
    if (n->wsplit) (void)hipFree(n->wsplit);
''')
open(p,'w').write(s)