/*
    enoki/array_router.h -- kept for source compatibility: in this implementation the contents of the reference's array_router.h
    (traits, routing, static arrays, structure support, ...) live in one header, enoki/array.h
*/
#pragma once
#include <enoki/array.h>
