/*
    enoki/array_math.h -- kept for source compatibility: the transcendental functions are part of enoki/array.h (device
    kernels / host packets), the polynomial helpers poly2 ... poly10 and the special functions of enoki/special.h
*/
#pragma once
#include <enoki/array.h>
#include <enoki/special.h>
