// pe_nl_fileformat/builtin_registry.h -- every model this build ships (reference: pe_nl_fileformat/builtin_registry.h:81-169; the
// Verilog module / port models and BSIM3v3.2 are not part of this build: a container naming them fails to load with
// `unsupported: no codec for model`, like any unregistered user model does in the reference).
#pragma once
#include <phy_engine/phy_engine.h>

#include "model_registry.h"

namespace phy_engine::pe_nl_fileformat
{
    inline model_registry const& default_registry()
    {
        static model_registry const reg = []
        {
            namespace pm = ::phy_engine::model;
            model_registry r{};
            auto all = [&]<class... M>() { (r.add(details::make_entry<M>()), ...); };
            // controller / mixed signal
            all.template operator()<pm::comparator, pm::relay, pm::single_pole_switch>();
            // digital blocks
            all.template operator()<pm::COUNTER4, pm::DFF, pm::DFF_ARSTN, pm::DLATCH, pm::FULL_ADDER, pm::FULL_SUB, pm::HALF_ADDER, pm::HALF_SUB, pm::JKFF, pm::MUL2,
                                    pm::RANDOM_GENERATOR4, pm::T_BAR_FF, pm::TFF>();
            all.template operator()<pm::AND, pm::CASE_EQ, pm::EIGHT_BIT_DISPLAY, pm::EIGHT_BIT_INPUT, pm::IMP, pm::INPUT, pm::IS_UNKNOWN, pm::NAND, pm::NIMP, pm::NOR, pm::NOT,
                                    pm::OR, pm::OUTPUT, pm::RESOLVE2, pm::SCHMITT_TRIGGER, pm::TICK_DELAY, pm::TRI, pm::XNOR, pm::XOR, pm::YES>();
            // generators, linear, non-linear
            all.template operator()<pm::pulse_gen, pm::sawtooth_gen, pm::square_gen, pm::triangle_gen>();
            all.template operator()<pm::CCCS, pm::CCVS, pm::IAC, pm::IDC, pm::VAC, pm::VCCS, pm::VCVS, pm::VDC, pm::capacitor, pm::coupled_inductors, pm::inductor, pm::op_amp,
                                    pm::resistance, pm::transformer, pm::transformer_center_tap>();
            all.template operator()<pm::BJT_NPN, pm::BJT_PNP, pm::PN_junction, pm::full_bridge_rectifier, pm::nmosfet, pm::pmosfet>();
            return r;
        }();
        return reg;
    }
}  // namespace phy_engine::pe_nl_fileformat
