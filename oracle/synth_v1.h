/* synth_v1 clip generator -- test infrastructure. */
#ifndef SYNTH_V1_H
#define SYNTH_V1_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Write one I420 frame (w*h*3/2 bytes, no padding) of clip time t into dst. */
void synth_v1_frame(uint8_t *dst, int w, int h, int t, uint32_t seed);
#ifdef __cplusplus
}
#endif
#endif
