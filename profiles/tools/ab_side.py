"""(History) same-box A/B of weight gradients on a second stream, round 3: with Engine.SIDE_PIXELS (removed again) the
small-grid layers took the separate BatchNorm-backward apply pass and their weight gradient + slab reduction ran on a side
stream beside the data-gradient chain.  MI355X, unet.py train step, b16 at 384 x 384, hipGraph replay, same box:

    side_pixels       0:  0 side ops, 6.678 ms/step = 2396 img/s
    side_pixels    9216: 16 side ops, 6.928 ms/step = 2310 img/s     (12 x 12 and 24 x 24 levels)
    side_pixels   36864: 26 side ops, 6.940 ms/step = 2306 img/s     (+ 48 x 48)
    side_pixels  147456: 46 side ops, 6.992 ms/step = 2288 img/s     (+ 96 x 96)

Every fork / join edge of the captured graph costs more than the overlapped 20-70 workgroup kernels gain; removed."""
