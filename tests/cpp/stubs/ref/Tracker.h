// TEST STUB — see System.h next to this file.
#pragma once
#include "System.h"
