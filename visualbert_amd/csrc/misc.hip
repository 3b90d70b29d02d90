extern "C" const char* vb_version(void) {
#ifdef VB_EMU
    return "visualbert_hip EMULATOR (developer tool, not the product) r1";
#else
    return "visualbert_hip gfx950 r1";
#endif
}
