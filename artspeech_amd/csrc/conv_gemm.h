#pragma once
#include "artspeech_hip.h"
