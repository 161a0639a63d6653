#!/bin/bash
# measurement: what the span screen's time is made of (compile-time -DLDW_SCREEN_EXP: 1 no table path, 2 no multi-cell path, 3 neither; results
# are wrong on purpose, the redo path repairs them): serial kernel stats per variant, built on the box (tools/r04_screen_variants.sh)
exec bash "$(dirname "$0")/r04_screen_variants.sh" base= notab=-DLDW_SCREEN_EXP=1 nocell=-DLDW_SCREEN_EXP=2 neither=-DLDW_SCREEN_EXP=3
