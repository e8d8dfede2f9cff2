// rtx_sources.cpp — the text of the device sources inside librtx.so, for the run-time compilation
// of scene-specialised kernels (rtx_special.cpp): the assembler copies the files in as they lie
// beside this one when the library is built.
#define RTX_EMBED(symbol, path)                           \
    asm(".section .rodata\n"                              \
        ".global " #symbol "\n" #symbol ":\n"             \
        ".incbin \"" path "\"\n"                          \
        ".byte 0\n"                                       \
        ".previous\n")
RTX_EMBED(rtx_src_core, "rtx_core.h");
RTX_EMBED(rtx_src_device, "rtx_device.h");
RTX_EMBED(rtx_src_det, "../../include/rtx_det.h");
RTX_EMBED(rtx_src_abi, "../../include/rtx.h");
