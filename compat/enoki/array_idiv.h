/*
    enoki/array_idiv.h -- kept for source compatibility: in this implementation the contents of the reference's array_idiv.h
    (traits, routing, static arrays, structure support, ...) live in one header, enoki/array.h
*/
#pragma once
#include <enoki/array.h>
