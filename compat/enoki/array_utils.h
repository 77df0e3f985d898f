/*
    enoki/array_utils.h -- kept for source compatibility: in this implementation the contents of the reference's array_utils.h
    (traits, routing, static arrays, structure support, ...) live in one header, enoki/array.h
*/
#pragma once
#include <enoki/array.h>
