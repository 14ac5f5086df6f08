graph [
  directed 1
  node [
    id 0
    label "0"
  ]
  node [
    id 1
    label "1"
  ]
  node [
    id 2
    label "2"
  ]
  node [
    id 3
    label "3"
  ]
  node [
    id 4
    label "4"
  ]
  node [
    id 5
    label "5"
  ]
  node [
    id 6
    label "6"
  ]
  node [
    id 7
    label "7"
  ]
  node [
    id 8
    label "8"
  ]
  edge [
    source 0
    target 1
    capacity 4412
  ]
  edge [
    source 0
    target 8
    capacity 6988
  ]
  edge [
    source 1
    target 0
    capacity 4412
  ]
  edge [
    source 1
    target 2
    capacity 5135
  ]
  edge [
    source 1
    target 4
    capacity 4180
  ]
  edge [
    source 2
    target 1
    capacity 5135
  ]
  edge [
    source 2
    target 3
    capacity 3250
  ]
  edge [
    source 3
    target 2
    capacity 3250
  ]
  edge [
    source 3
    target 4
    capacity 2320
  ]
  edge [
    source 3
    target 8
    capacity 5285
  ]
  edge [
    source 3
    target 6
    capacity 2014
  ]
  edge [
    source 4
    target 1
    capacity 4180
  ]
  edge [
    source 4
    target 3
    capacity 2320
  ]
  edge [
    source 4
    target 5
    capacity 2488
  ]
  edge [
    source 5
    target 4
    capacity 2488
  ]
  edge [
    source 5
    target 6
    capacity 4147
  ]
  edge [
    source 6
    target 3
    capacity 2014
  ]
  edge [
    source 6
    target 5
    capacity 4147
  ]
  edge [
    source 6
    target 7
    capacity 4748
  ]
  edge [
    source 7
    target 6
    capacity 4748
  ]
  edge [
    source 7
    target 8
    capacity 9978
  ]
  edge [
    source 8
    target 0
    capacity 6988
  ]
  edge [
    source 8
    target 3
    capacity 5285
  ]
  edge [
    source 8
    target 7
    capacity 9978
  ]
]
